"""GPU experiment (round-5 verdict item 1a): the bench batch (4096 stereo streams x K blocks of 2048) encoded - and decoded -
as G staggered stream groups: G encoder / decoder objects of 4096/G contiguous streams, each driven on its own HIP stream, the
calls enqueued back to back, joined with events.  Streams are independent (ulcEncoder.c:93-158: all state is per
ULC_EncoderState_t), so the bytes must equal the single object's: the md5 over slots + sizes is printed.
GPU_MAX_HW_QUEUES must be in the environment before HIP initialises (one encoder already uses four streams).
  usage: [GPU_MAX_HW_QUEUES=8] python tools/enc_groups.py G [K] [iters] [stagger]"""
import os, sys, hashlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd")); sys.path.insert(0, ROOT)
import ulc_amd, bench
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
B = 4096
BS, C = 2048, 2
pcm = bench.make_pcm(torch, B, K * BS, dev, seed=1234)
cuts = [B * g // G for g in range(G + 1)]
encs = [ulc_amd.BatchEncoder(cuts[g + 1] - cuts[g], C, BS, 44100, K) for g in range(G)]
decs = [ulc_amd.BatchDecoder(cuts[g + 1] - cuts[g], C, BS, K) for g in range(G)]
slot = encs[0].slot
d_out = torch.zeros(B * K * slot, dtype=torch.uint8, device=dev); d_bits = torch.zeros(B * K, dtype=torch.int32, device=dev)
d_dec = torch.zeros(B * K * BS * C, dtype=torch.float32, device=dev); d_dbits = torch.zeros(B * K, dtype=torch.int32, device=dev)
for e in encs: e.set_timing(False)
for d in decs: d.set_timing(False)
main = torch.cuda.current_stream(dev)
streams = [main] + [torch.cuda.Stream(dev) for _ in range(G - 1)]
evs = [torch.cuda.Event() for _ in range(G)]


def fork():
    e = torch.cuda.Event(); e.record(main)
    for s in streams[1:]: s.wait_event(e)


def join():
    for g in range(1, G):
        evs[g].record(streams[g]); main.wait_event(evs[g])


def encode_all():
    fork()
    for g in range(G):
        s0 = cuts[g]
        encs[g].encode_dev(pcm.data_ptr() + s0 * K * BS * C * 4, K, d_out.data_ptr() + s0 * K * slot, d_bits.data_ptr() + s0 * K * 4,
                           p0=50.0, stream=streams[g].cuda_stream)
    join()


def decode_all():
    fork()
    for g in range(G):
        s0 = cuts[g]
        decs[g].decode_dev(d_out.data_ptr() + s0 * K * slot, slot, K, d_dec.data_ptr() + s0 * K * BS * C * 4, d_dbits.data_ptr() + s0 * K * 4,
                           stream=streams[g].cuda_stream)
    join()


e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
te = td = 0.0; n = 0
for it in range(ITERS):
    e0.record(main); encode_all(); e1.record(main); decode_all(); e2.record(main); torch.cuda.synchronize()
    if it >= 3:
        te += e0.elapsed_time(e1); td += e1.elapsed_time(e2); n += 1
md5 = hashlib.md5(d_out.cpu().numpy().tobytes() + d_bits.cpu().numpy().tobytes()).hexdigest()[:8]
md5d = hashlib.md5(d_dec.cpu().numpy().tobytes() + d_dbits.cpu().numpy().tobytes()).hexdigest()[:8]
print("G=%d K=%d queues=%s | enc %.3f dec %.3f step %.3f ms | md5 enc %s dec %s" % (G, K, os.environ.get("GPU_MAX_HW_QUEUES", "default"), te / n, td / n, (te + td) / n, md5, md5d))
