#!/usr/bin/env python3
"""bench.py — whole-job throughput of the batched ulc-codec hot path on MI355X.

Workload (BASELINE.json configs[1] + configs[2] shape): B independent 44.1 kHz stereo
streams, BlockSize 2048, VBR quality 50.  One *step* = encode K consecutive blocks of
every stream (ulcx_encode_dev) and decode the B*K blocks just produced
(ulcx_decode_dev); inputs, outputs and codec state stay resident in HBM.
value = channel-samples that went through encode+decode per second (Msamples/s).

Prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel,
priced live from hipEvents on the launch stream) and, at N=1, `cpu_baseline` (the C
oracle timed on the host cores on a bounded sample of the same workload).
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BS, CH, RATE, QUALITY = 2048, 2, 44100, 50.0


def make_pcm(torch, B, n, device, seed):
    """Seeded synthetic PCM16-grid audio: 3 tones + noise + sparse decaying bursts; second
    channel = delayed scaled copy + independent noise."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.arange(n, device=device, dtype=torch.float32)[None, :] / RATE
    x = torch.zeros(B, n, device=device)
    for _ in range(3):
        f = torch.exp(torch.empty(B, 1, device=device).uniform_(4.382, 9.393, generator=g))      # 80 Hz .. 12 kHz
        a = torch.empty(B, 1, device=device).uniform_(0.05, 0.3, generator=g)
        ph = torch.empty(B, 1, device=device).uniform_(0, 6.2831853, generator=g)
        x += a * torch.sin(6.2831853 * f * t + ph)
    x += 0.02 * torch.randn(B, n, device=device, generator=g)
    # bursts: ~4 per second at random positions, exponential decay tau = 300 samples
    nb = max(1, int(4 * n / RATE))
    pos = torch.randint(0, n, (B, nb), device=device, generator=g)
    amp = 10 ** torch.empty(B, nb, device=device).uniform_(-2.5, -0.5, generator=g)
    idx = torch.arange(n, device=device)[None, :]
    for j in range(nb):
        d = (idx - pos[:, j:j + 1]).float()
        env = torch.where(d >= 0, torch.exp(-d.clamp(min=0) / 300.0), torch.zeros_like(d)) * amp[:, j:j + 1]
        x += env * torch.randn(B, n, device=device, generator=g)
    y = 0.8 * torch.roll(x, 7, dims=1) + 0.01 * torch.randn(B, n, device=device, generator=g)
    pcm = torch.stack([x, y], dim=2)
    pcm = torch.clamp(torch.round(pcm * 32768.0), -32768, 32767) * (1.0 / 32768.0)
    return pcm.contiguous()          # [B][n][2] f32


def cpu_baseline(sample_pcm, n_blocks, target_seconds=4.0):
    """Oracle (kind 'port') encode+decode on the host cores.  sample_pcm: numpy [S][n][C]."""
    import numpy as np
    from ulc_testlib import oracle, ptr, f32p, u8p, i32p
    lib = oracle()
    S = sample_pcm.shape[0]
    slot = 2 * CH * BS + 16
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))

    def one(pcm, out, bits, dec):
        lib.orc_encode_stream_vbr(RATE, CH, BS, ptr(pcm, f32p), n_blocks, QUALITY, ptr(out, u8p), slot, ptr(bits, i32p), None, None)
        lib.orc_decode_stream(CH, BS, ptr(out, u8p), slot, n_blocks, ptr(dec, f32p), None)

    flat = [np.ascontiguousarray(sample_pcm[i].reshape(-1)) for i in range(S)]
    # warm the oracle's lazily built tables single-threaded, and time one stream to size the sample
    out0 = np.zeros((n_blocks, slot), np.uint8); b0 = np.zeros(n_blocks, np.int32); d0 = np.zeros(n_blocks * BS * CH, np.float32)
    one(flat[0], out0, b0, d0)
    t0 = time.perf_counter(); one(flat[0], out0, b0, d0); t1 = time.perf_counter() - t0
    reps = max(1, int(target_seconds / max(t1, 1e-4)))
    done = [0] * threads

    def worker(i):
        out = np.zeros((n_blocks, slot), np.uint8); b = np.zeros(n_blocks, np.int32); d = np.zeros(n_blocks * BS * CH, np.float32)
        for r in range(reps):
            one(flat[(i + r) % S], out, b, d)
            done[i] += 1

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    el = time.perf_counter() - t0
    streams = sum(done)
    samples = streams * n_blocks * BS * CH
    return {"value": samples / el / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{streams} stream-encodes+decodes of {n_blocks} blocks ({S} distinct seeded streams of the bench batch) in {el:.1f} s, "
                      f"one oracle instance per thread, gcc -O2 scalar C"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=4096, help="independent streams per GPU (BASELINE configs[1]: 4096)")
    ap.add_argument("--blocks", type=int, default=16, help="consecutive blocks per stream per step")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pcm16", action="store_true", help="separate configuration (SURVEY.md 8f rank 4): PCM16 ingest and PCM16 output "
                    "fused into the first/last kernel instead of the C API's f32; NOT the headline line")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    cpu = None
    import numpy as np
    import torch
    import ulc_amd
    if not os.path.exists(ulc_amd.LIB_PATH):
        raise SystemExit("libulc_amd.so missing — run __graft_entry__.build(); there is no CPU fallback")
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    B, K = args.streams, args.blocks
    n = K * BS
    # independent streams shard across ranks by plain batch split (shard.py): rank r owns global streams
    # [r*B, (r+1)*B) — per-GPU work fixed as N grows (weak scaling), no collective on the data path
    import shard
    ids = shard.weak_scaling_ids(B, rank)
    pcm = make_pcm(torch, B, n, dev, seed=1234 + ids[0])
    enc = ulc_amd.BatchEncoder(B, CH, BS, RATE, K, device=dev.index)
    dec = ulc_amd.BatchDecoder(B, CH, BS, K, device=dev.index)
    slot = enc.slot
    d_out = torch.zeros(B * K * slot, dtype=torch.uint8, device=dev)
    d_bits = torch.zeros(B * K, dtype=torch.int32, device=dev)
    d_dec = torch.zeros(B * n * CH, dtype=torch.float32, device=dev)
    d_dbits = torch.zeros(B * K, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    if args.pcm16:
        pcm16 = torch.clamp(torch.round(pcm * 32767.0), -32768, 32767).to(torch.int16)
        d_dec16 = torch.zeros(B * n * CH, dtype=torch.int16, device=dev)
        del pcm
        pcm = pcm16.to(torch.float32) * (2.0 ** -15)            # what the CPU baseline leg would read

    def step16():
        enc.encode_dev_pcm16(pcm16.data_ptr(), K, d_out.data_ptr(), d_bits.data_ptr(), mode=ulc_amd.MODE_VBR, p0=QUALITY, stream=stream)
        dec.decode_dev_pcm16(d_out.data_ptr(), slot, K, d_dec16.data_ptr(), d_dbits.data_ptr(), stream=stream)

    def step():
        if args.pcm16:
            return step16()
        enc.encode_dev(pcm.data_ptr(), K, d_out.data_ptr(), d_bits.data_ptr(), mode=ulc_amd.MODE_VBR, p0=QUALITY, stream=stream)
        dec.decode_dev(d_out.data_ptr(), slot, K, d_dec.data_ptr(), d_dbits.data_ptr(), stream=stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    enc_ms, dec_ms = {}, {}
    t0 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
    enc_t = dec_t = 0.0
    for i in range(args.steps):
        step()
    barrier()
    el = time.perf_counter() - t0
    # per-kernel device times of one more (untimed) step, from the library's own hipEvents on the launch stream
    acc_enc, acc_dec, nacc = {}, {}, 3
    for _ in range(nacc):
        step(); torch.cuda.synchronize(dev)
        for k_, v in enc.stage_ms().items(): acc_enc[k_] = acc_enc.get(k_, 0.0) + v / nacc
        for k_, v in dec.stage_ms().items(): acc_dec[k_] = acc_dec.get(k_, 0.0) + v / nacc
    el = shard.max_over_ranks(el, dist, dev)
    bits_host = d_bits.cpu().numpy()
    dbits_host = d_dbits.cpu().numpy()
    ok = bool((dbits_host > 0).all() and (dbits_host <= bits_host).all())
    value = shard.whole_job_throughput(B * K * BS * CH * args.steps, world, el) / 1e6

    # ---- roofline of the dominant kernel (algorithmic bytes: SURVEY.md §8d / BASELINE.md §4)
    mean_bytes = float(bits_host.mean()) / 8.0
    smp_bytes = 2 if args.pcm16 else 4
    alg_bytes_block = smp_bytes * CH * BS + mean_bytes + 8  # f32 (PCM16 with --pcm16) in (or out) + stream bytes + size/WindowCtrl metadata
    pseudo = ("cbr_probe_passes", "k_heapsel", "wc_pipeline_exposed")   # intervals, not kernels (join wait / side-stream launches)
    allk = {**{("enc", k_): v for k_, v in acc_enc.items() if k_ not in pseudo}, **{("dec", k_): v for k_, v in acc_dec.items()}}
    (side, kname), kms = max(allk.items(), key=lambda kv: kv[1])
    # kms is the kernel's time per step; a kernel launched n times per step (k_xf: one launch per chunk of blocks of
    # the window-control pipeline) has kms = sum of its n launches, each timed by its own hipEvent pair on its stream.
    # Per launch: bytes/n over kms/n - the same ratio.  profiles/*_pmc_summary.json carries n and the measured traffic.
    launches = float(enc.xf_launches()) if kname == "k_xf" else 1.0      # from the library: chunks of the last call
    traffic = None
    pj = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if os.path.exists(pj):
        try:
            ent = json.load(open(pj)).get(kname, {})
            traffic = ent.get("hbm_bytes_per_launch")                   # (key name: it is the kernel's bytes per STEP of 65536 blocks)
            if traffic is not None:
                traffic = traffic * (B * K / 65536.0) / launches
        except Exception:
            traffic = None
    launch_bytes = alg_bytes_block * B * K / launches
    kms_launch = kms / launches
    achieved = launch_bytes / (kms_launch * 1e-3) / 1e9
    enc_total = sum(acc_enc.values()); dec_total = sum(acc_dec.values())

    if rank == 0 and world == 1 and not args.no_cpu:
        S = 8
        sample = pcm[:S].cpu().numpy()
        # release the GPU objects' host threads are idle; oracle runs on host cores only
        cpu = cpu_baseline(sample, K)

    if rank == 0:
        line = {
            "metric": "encode+decode Msamples/s at BlockSize=2048 stereo (channel-samples through VBR-50 encode then decode)",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Batch={B} independent 44.1 kHz stereo streams/GPU x {K} blocks, BlockSize=2048, VBR -50 encode "
                                   f"(BASELINE configs[1]) then decode of the {B*K} blocks produced (configs[2] shape)"
                                   + (" -- PCM16 ingest/output variant (int16 samples in HBM, not the C API's f32)" if args.pcm16 else ""),
                       "streams_per_gpu": B, "blocks_per_stream_per_step": K, "block_size": BS, "channels": CH, "rate_hz": RATE,
                       "parallelism": f"batch split over {world} GPU(s), no collective on the data path"},
            "roofline": {"bound": "hbm", "kernel": f"{kname} ({side})", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_block": alg_bytes_block, "blocks_per_launch": B * K / launches,
                         "launches_per_step": launches, "kernel_ms": kms_launch},
            "whole_pipeline": {"encode_ms": enc_total, "decode_ms": dec_total,
                               "encode_Msamples_s": B * K * BS * CH / (enc_total * 1e-3) / 1e6,
                               "decode_Msamples_s": B * K * BS * CH / (dec_total * 1e-3) / 1e6,
                               "encode_hbm_frac": alg_bytes_block * B * K / (enc_total * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "decode_hbm_frac": alg_bytes_block * B * K / (dec_total * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "mean_block_bytes": mean_bytes, "decode_ok": ok},
            "kernels_ms": {**{f"enc.{k_}": round(v, 4) for k_, v in acc_enc.items()}, **{f"dec.{k_}": round(v, 4) for k_, v in acc_dec.items()}},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
