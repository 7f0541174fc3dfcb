#!/usr/bin/env python3
"""bench.py — whole-job throughput of the batched ulc-codec hot path on MI355X.

Workloads (BASELINE.json `configs`; --config picks one, the default is the headline):
  vbr50         configs[1] + configs[2]: B independent 44.1 kHz stereo streams per GPU, BlockSize 2048, VBR quality 50;
                the decode leg decodes the B*K blocks the encode leg produced.  Weak scaling (4096 streams per GPU).
  cbr64_48k     configs[3]: CBR 64 kbps, 48 kHz M/S stereo, BlockSize 2048, 32768 streams in total, split over the GPUs
                (fixed total: strong scaling).
  wswitch_4096  configs[4]: window-switch stress, transient-heavy stereo (>= 5 bursts/s, onset strengths over three
                decades), BlockSize 4096, VBR 50, 16384 streams in total, split over the GPUs (strong scaling).
--mode both|encode|decode selects what a *step* is: encode K consecutive blocks of every stream (ulcx_encode_dev), decode
them (ulcx_decode_dev), or both back to back.  Inputs, outputs and codec state stay resident in HBM; the same K input
blocks are fed every step (the codec state advances, the data repeats - throughput does not depend on it).
value = channel-samples that went through the step per second (Msamples/s; a sample counts once, whichever legs the step has),
whole job over all ranks.

Prints ONE JSON line (contract in the task statement) with `roofline` (the dominant kernel priced live from hipEvents on
its launch stream, plus the whole encode / decode legs against the same algorithmic bytes) and, at N=1, `cpu_baseline`
(the C oracle timed on the host cores on a bounded sample of the same workload).
The K timed steps run without the library's per-kernel hipEvents (22 marker packets per encode call, 3 per decode: 0.09 ms
per step; ULCX_BENCH_TIMING=1 keeps them in); the per-kernel times and the roofline's kernel_ms come from three more steps
of the same workload run straight after the timed region with the events on.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BS, CH, RATE, QUALITY = 2048, 2, 44100, 50.0          # the headline workload (tests import these and make_pcm)

CONFIGS = {
    #               BlockSize rate   mode  param  streams/GPU  total streams  bursts/s  decades  BASELINE.json
    # blocks: consecutive blocks per stream per step.  vbr50: 32, the figure SURVEY.md 8(d) writes for config 2 ("4096 stereo streams
    # x (>= 32 blocks each)"; rounds 1-2 ran 16: `--blocks 16` reproduces those lines); the fixed-total configurations keep 16.
    "vbr50":        dict(bs=2048, rate=44100, mode="vbr", p0=50.0, per_gpu=4096, total=None,  bursts=4.0, decades=2.0, blocks=32, ref="configs[1] encode, configs[2] decode"),
    "cbr64_48k":    dict(bs=2048, rate=48000, mode="cbr", p0=64.0, per_gpu=None, total=32768, bursts=4.0, decades=2.0, blocks=16, ref="configs[3]"),
    "wswitch_4096": dict(bs=4096, rate=44100, mode="vbr", p0=50.0, per_gpu=None, total=16384, bursts=6.0, decades=3.0, blocks=16, ref="configs[4]"),
}


def make_pcm(torch, B, n, device, seed, bursts_per_s=4.0, decades=2.0):
    """Seeded synthetic PCM16-grid audio: 3 tones + noise + sparse decaying bursts; second
    channel = delayed scaled copy + independent noise.  bursts_per_s / decades: rate of the noise bursts
    and the spread of their onset strengths (config 5: >= 5 per second over three decades)."""
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
    # bursts at random positions, exponential decay tau = 300 samples
    nb = max(1, int(bursts_per_s * n / RATE))
    pos = torch.randint(0, n, (B, nb), device=device, generator=g)
    amp = 10 ** torch.empty(B, nb, device=device).uniform_(-0.5 - decades, -0.5, generator=g)
    idx = torch.arange(n, device=device)[None, :]
    for j in range(nb):
        d = (idx - pos[:, j:j + 1]).float()
        env = torch.where(d >= 0, torch.exp(-d.clamp(min=0) / 300.0), torch.zeros_like(d)) * amp[:, j:j + 1]
        x += env * torch.randn(B, n, device=device, generator=g)
    y = 0.8 * torch.roll(x, 7, dims=1) + 0.01 * torch.randn(B, n, device=device, generator=g)
    pcm = torch.stack([x, y], dim=2)
    pcm = torch.clamp(torch.round(pcm * 32768.0), -32768, 32767) * (1.0 / 32768.0)
    return pcm.contiguous()          # [B][n][2] f32


def host_cpu_budget():
    """(threads to use, facts): the CPUs this process may run on = min(scheduler affinity, cgroup CPU quota)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                                     # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:                                                 # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            quota = None
    model, phys = "unknown", set()
    try:
        pid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                phys.add((pid, ln.split(":", 1)[1].strip()))
    except Exception:
        pass
    threads = aff if quota is None else max(1, min(aff, int(quota + 0.999)))
    return threads, {"affinity_cpus": aff, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count(),
                     "physical_cores_in_cpuinfo": len(phys) or None, "cpu_model": model}


def cpu_baseline(sample_pcm, n_blocks, cfg, legs, target_seconds=8.0):
    """Oracle (kind 'port') on the host cores: the legs of the step on a bounded sample.  sample_pcm: numpy [S][n][C].
    The worker loop is C (oracle/orc_bench.c: one POSIX thread per CPU this process may use, each with its own oracle
    encoder / decoder, time-boxed); Python only prepares the sample and reads the counts."""
    import ctypes as C
    import numpy as np
    from ulc_testlib import oracle, ptr, f32p, u8p, i32p
    lib = oracle()
    lib.orc_bench_threads.restype = C.c_int
    lib.orc_bench_threads.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, u8p, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    S = sample_pcm.shape[0]
    bs, rate = cfg["bs"], cfg["rate"]
    slot = 2 * CH * bs + 16
    threads, facts = host_cpu_budget()
    mode = 0 if cfg["mode"] == "vbr" else 1
    enc_fn = lib.orc_encode_stream_vbr if mode == 0 else lib.orc_encode_stream_cbr
    legbits = (1 if ("encode" in legs or "decode" not in legs) else 0) | (2 if "decode" in legs else 0)
    flat = np.ascontiguousarray(sample_pcm.reshape(S, -1).astype(np.float32))
    outs = np.zeros((S, n_blocks, slot), np.uint8)
    for i in range(S):                                   # (a decode-only leg needs encoded input: made once, untimed)
        b = np.zeros(n_blocks, np.int32)
        enc_fn(rate, CH, bs, ptr(flat[i], f32p), n_blocks, cfg["p0"], ptr(outs[i], u8p), slot, ptr(b, i32p), None, None)

    def run(nthr, seconds):
        el, n = C.c_double(0.0), C.c_longlong(0)
        rc = lib.orc_bench_threads(nthr, mode, legbits, rate, CH, bs, ptr(flat, f32p), ptr(outs, u8p), S, n_blocks, slot, cfg["p0"], seconds,
                                   C.byref(el), C.byref(n))
        if rc != 0:
            raise RuntimeError(f"orc_bench_threads failed: {rc}")
        return n.value, el.value

    n1, el1 = run(1, 2.0)                                # one thread alone: what a single core does
    one_thread = n1 * n_blocks * bs * CH / el1 / 1e6
    streams, el = run(threads, target_seconds)
    value = streams * n_blocks * bs * CH / el / 1e6
    # if the full count scales badly, half of it tells SMT siblings / a quota from a harness limit (there is none: the loop is C)
    half = None
    if threads >= 4:
        nh, elh = run(threads // 2, min(target_seconds, 4.0))
        half = {"threads": threads // 2, "value": nh * n_blocks * bs * CH / elh / 1e6}
    return {"value": value, "unit": "Msamples/s", "cores": threads, "kind": "port", "one_thread": one_thread,
            "scaling_per_core": value / (one_thread * threads), "half_the_threads": half, **facts,
            "sample": f"{streams} stream-{'+'.join(legs)}s of {n_blocks} blocks ({S} distinct seeded streams of the bench batch) in {el:.1f} s on "
                      f"{threads} POSIX threads (C worker loop, oracle/orc_bench.c; one oracle encoder/decoder per thread); scalar C port of the "
                      f"reference built gcc -O2 -ffp-contract=off as the reference's Makefile builds libulc - the reference's own SIMD lives in "
                      f"libfourier, which is absent from the tree, so no AVX2/FMA reference path can be timed (an -mavx2 build of the port "
                      f"without contraction is bit-identical and within a few per cent: DESIGN.md §8)"}


def run_secondary(torch, ulc_amd, dev, name, B, K, steps=3):
    """One short run of another BASELINE configuration at one GPU's share (outside the headline's timed region): so that a
    regression in the rate search (cbr64_48k) or in the BlockSize-4096 / window-switching kernels (wswitch_4096) shows in the
    driver's line.  A step = encode K blocks of B streams, then decode them; wall clock around `steps` steps after one warm-up,
    the legs from the library's own hipEvents of one more step."""
    global RATE
    cfg = CONFIGS[name]
    bs, rate = cfg["bs"], cfg["rate"]
    keep_rate, RATE = RATE, rate
    try:
        pcm = make_pcm(torch, B, K * bs, dev, seed=4321, bursts_per_s=cfg["bursts"], decades=cfg["decades"])
    finally:
        RATE = keep_rate
    enc = ulc_amd.BatchEncoder(B, CH, bs, rate, K, device=dev.index)
    dec = ulc_amd.BatchDecoder(B, CH, bs, K, device=dev.index)
    slot = enc.slot
    d_out = torch.zeros(B * K * slot, dtype=torch.uint8, device=dev); d_bits = torch.zeros(B * K, dtype=torch.int32, device=dev)
    d_dec = torch.zeros(B * K * bs * CH, dtype=torch.float32, device=dev); d_dbits = torch.zeros(B * K, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    emode = ulc_amd.MODE_VBR if cfg["mode"] == "vbr" else ulc_amd.MODE_CBR

    def step():
        enc.encode_dev(pcm.data_ptr(), K, d_out.data_ptr(), d_bits.data_ptr(), mode=emode, p0=cfg["p0"], stream=stream)
        dec.decode_dev(d_out.data_ptr(), slot, K, d_dec.data_ptr(), d_dbits.data_ptr(), stream=stream)

    enc.set_timing(False); dec.set_timing(False)
    step(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / steps * 1e3
    enc.set_timing(True); dec.set_timing(True)
    step(); torch.cuda.synchronize(dev)
    enc_ms = sum(enc.stage_ms().values()); dec_ms = sum(dec.stage_ms().values())
    bits_h = d_bits.cpu().numpy(); dbits_h = d_dbits.cpu().numpy()
    ok = bool((dbits_h > 0).all() and (dbits_h <= bits_h).all())
    res = {"workload": f"{name} ({cfg['ref']}) at one GPU's share: {B} streams x {K} blocks, BlockSize={bs}, "
                       + ("VBR -%g" % cfg["p0"] if cfg["mode"] == "vbr" else "CBR %g kbps" % cfg["p0"]) + f", {rate / 1000:g} kHz stereo",
           "steps": steps, "ms_per_step": ms, "encode_ms": enc_ms, "decode_ms": dec_ms, "decode_ok": ok,
           "Msamples_s": B * K * bs * CH / (ms * 1e-3) / 1e6, "mean_block_bytes": float(bits_h.mean()) / 8.0}
    if cfg["mode"] != "vbr":
        budget = int((bs * cfg["p0"]) * 1000.0 / rate)
        res["cbr_budget_bits"] = budget
        res["cbr_max_block_bits"] = int(bits_h.max())
    enc.close(); dec.close()
    return res


def self_launch(n):
    """Start n ranks of this script (one per GPU) and wait for them.  The launcher process initialises no GPU runtime (a
    process that has must not be replaced or forked); rank r gets LOCAL_RANK = r and talks to the others over 127.0.0.1."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(None if r == 0 else subprocess.DEVNULL)))
    worst = 0
    deadline = None
    while procs:
        for p in list(procs):
            rc = p.poll()
            if rc is None:
                continue
            procs.remove(p)
            if rc != 0:
                worst = worst or rc
                if deadline is None:                      # one rank failed: the others would wait for it in a collective
                    deadline = time.time() + 20.0
        if deadline is not None and time.time() > deadline:
            for p in procs:
                p.kill()                                  # (exact children of this process, by handle)
            for p in procs:
                p.wait()
            procs = []
        time.sleep(0.05)
    if worst:
        sys.stderr.write(f"bench.py: a rank of the {n}-GPU run failed (exit code {worst}); no result line\n")
    return worst if worst > 0 else (1 if worst else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="vbr50")
    ap.add_argument("--mode", choices=["both", "encode", "decode"], default="both")
    ap.add_argument("--streams", type=int, default=0, help="independent streams per GPU (default: the config's; fixed-total configs split their total over the GPUs)")
    ap.add_argument("--blocks", type=int, default=0, help="consecutive blocks per stream per step (default: the config's: 32 for vbr50, 16 otherwise)")
    ap.add_argument("--total-streams", type=int, default=0, help="fixed-total (strong-scaling) configurations: the total split over the GPUs "
                    "(default: the config's: 32768 / 16384); a reduced total is a test run, never a result line for BASELINE's configuration")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other two configurations that the default "
                    "1-GPU headline run appends as `secondary` (outside the timed region)")
    ap.add_argument("--pmc-summary", default="", help="tools/pmc_summary.py output of a rocprofv3 --pmc run of THIS command: fills roofline.traffic "
                    "(default: the committed profiles/r06_pmc_summary.json when its _meta.src_rev equals the library's ulcx_build_rev() "
                    "and its workload is this one; null otherwise)")
    ap.add_argument("--pcm16", action="store_true", help="separate configuration (SURVEY.md 8f rank 4): PCM16 ingest and PCM16 output "
                    "fused into the first/last kernel instead of the C API's f32; NOT the headline line")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it: this process becomes the launcher.  It starts N
    # ranks of itself (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), never touches
    # the GPU, forwards rank 0's JSON line and exits with the worst exit code.  Under torch.distributed.run the
    # environment already carries WORLD_SIZE and this is skipped.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: they must agree "
                         f"(python bench.py --gpus N starts its own N ranks)")
    cfg = CONFIGS[args.config]
    bs, rate = cfg["bs"], cfg["rate"]
    global RATE
    RATE = rate

    cpu = None
    import numpy as np
    import torch
    import ulc_amd
    if not os.path.exists(ulc_amd.LIB_PATH):
        raise SystemExit("libulc_amd.so missing — run __graft_entry__.build(); there is no CPU fallback")
    # ULCX_BENCH_SHARE_GPU=1 (tests only, never a result): the ranks share the visible device(s) and talk over gloo, so the
    # N > 1 code path of this file can run on a 1-GPU box; the line it prints says so in `data`.
    share = world > 1 and os.environ.get("ULCX_BENCH_SHARE_GPU") == "1"
    ndev = torch.cuda.device_count()
    if ndev < (1 if share else world):
        raise SystemExit(f"bench.py: {world} ranks but only {ndev} visible GPU(s): one process per GPU")
    dist = None
    dev_index = (local_rank % ndev) if share else (local_rank if world > 1 else 0)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if share: dist.init_process_group("gloo")
        else: dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    import shard
    # independent streams shard across ranks by plain batch split (shard.py), no collective on the data path.
    # weak: every rank owns its own B streams; strong (fixed-total configs): rank r owns stream_range(total, r, world)
    strong = cfg["total"] is not None and not args.streams
    total_streams = (args.total_streams or cfg["total"]) if strong else None
    if strong:
        lo, hi = shard.stream_range(total_streams, rank, world)
        B, first_id = hi - lo, lo
    else:
        B = args.streams or cfg["per_gpu"] or 4096
        first_id = shard.weak_scaling_ids(B, rank)[0]
    K = args.blocks or cfg["blocks"]
    n = K * bs
    pcm = make_pcm(torch, B, n, dev, seed=1234 + first_id, bursts_per_s=cfg["bursts"], decades=cfg["decades"])
    enc = ulc_amd.BatchEncoder(B, CH, bs, rate, K, device=dev.index)
    dec = ulc_amd.BatchDecoder(B, CH, bs, K, device=dev.index)
    slot = enc.slot
    d_out = torch.zeros(B * K * slot, dtype=torch.uint8, device=dev)
    d_bits = torch.zeros(B * K, dtype=torch.int32, device=dev)
    d_dec = torch.zeros(B * n * CH, dtype=torch.float32, device=dev)
    d_dbits = torch.zeros(B * K, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    emode = ulc_amd.MODE_VBR if cfg["mode"] == "vbr" else ulc_amd.MODE_CBR
    legs = ["encode", "decode"] if args.mode == "both" else [args.mode]

    if args.pcm16:
        pcm16 = torch.clamp(torch.round(pcm * 32767.0), -32768, 32767).to(torch.int16)
        d_dec16 = torch.zeros(B * n * CH, dtype=torch.int16, device=dev)
        del pcm
        pcm = pcm16.to(torch.float32) * (2.0 ** -15)            # what the CPU baseline leg would read

    def do_encode():
        if args.pcm16:
            enc.encode_dev_pcm16(pcm16.data_ptr(), K, d_out.data_ptr(), d_bits.data_ptr(), mode=emode, p0=cfg["p0"], stream=stream)
        else:
            enc.encode_dev(pcm.data_ptr(), K, d_out.data_ptr(), d_bits.data_ptr(), mode=emode, p0=cfg["p0"], stream=stream)

    def do_decode():
        if args.pcm16:
            dec.decode_dev_pcm16(d_out.data_ptr(), slot, K, d_dec16.data_ptr(), d_dbits.data_ptr(), stream=stream)
        else:
            dec.decode_dev(d_out.data_ptr(), slot, K, d_dec.data_ptr(), d_dbits.data_ptr(), stream=stream)

    def step():
        if "encode" in legs: do_encode()
        if "decode" in legs: do_decode()

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.mode == "decode":
        do_encode()                                              # the blocks the decode leg reads: produced once, untimed
    # the library's per-kernel hipEvents are marker packets between kernels: off for the timed region (a caller that wants
    # throughput does not record them), on again for the per-kernel breakdown below; ULCX_BENCH_TIMING=1 keeps them on
    keep_ev = bool(os.environ.get("ULCX_BENCH_TIMING"))
    enc.set_timing(keep_ev); dec.set_timing(keep_ev)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    barrier()
    el_rank = time.perf_counter() - t0
    enc.set_timing(True); dec.set_timing(True)
    # per-kernel device times of a few more (untimed) steps, from the library's own hipEvents on the launch stream
    acc_enc, acc_dec, nacc = {}, {}, 3
    for _ in range(nacc):
        step(); torch.cuda.synchronize(dev)
        if "encode" in legs:
            for k_, v in enc.stage_ms().items(): acc_enc[k_] = acc_enc.get(k_, 0.0) + v / nacc
        if "decode" in legs:
            for k_, v in dec.stage_ms().items(): acc_dec[k_] = acc_dec.get(k_, 0.0) + v / nacc
    el = shard.max_over_ranks(el_rank, dist, None if share else dev)
    per_rank_ms = [el_rank / args.steps * 1e3]
    if dist is not None:
        got = [None] * world
        dist.all_gather_object(got, el_rank / args.steps * 1e3)
        per_rank_ms = got
    if args.mode == "encode":
        do_decode(); torch.cuda.synchronize(dev)                 # the check below wants a decode of the last encode either way
    bits_host = d_bits.cpu().numpy()
    dbits_host = d_dbits.cpu().numpy()
    ok = bool((dbits_host > 0).all() and (dbits_host <= bits_host).all())
    assert ok, "decode of the encoded blocks failed: some block was rejected or consumed more bits than were written"
    units = B * K * bs * CH                                      # channel-samples through the step (each goes through every leg of it), this rank
    if strong:
        total_units = total_streams * K * bs * CH
        value = total_units * args.steps / el / 1e6
    else:
        value = shard.whole_job_throughput(units * args.steps, world, el) / 1e6

    # ---- roofline (algorithmic bytes: SURVEY.md §8d / BASELINE.md §4)
    mean_bytes = float(bits_host.mean()) / 8.0
    smp_bytes = 2 if args.pcm16 else 4
    alg_bytes_block = smp_bytes * CH * bs + mean_bytes + 8      # samples in (or out) + stream bytes + size/WindowCtrl metadata
    pseudo = ("cbr_probe_passes", "k_heapsel", "wc_pipeline_exposed")   # intervals, not kernels (join wait / side-stream launches)
    allk = {**{("enc", k_): v for k_, v in acc_enc.items() if k_ not in pseudo}, **{("dec", k_): v for k_, v in acc_dec.items()}}
    (side, kname), kms = max(allk.items(), key=lambda kv: kv[1])
    # A stage interval of the main stream also holds what the side streams run beside it (k_select's holds the noise
    # chain's kernels: 1.3 ms in-stream, 0.74 ms alone), so the largest interval is not always the largest kernel.  For the
    # one-pass (VBR) configurations the roofline kernel is fixed: the transform for encode / both (largest kernel by its
    # own time and by HBM traffic, each launch timed by its own event pair), the synthesis kernel for decode.
    if cfg["mode"] == "vbr":
        want = ("dec", "k_dsyn") if args.mode == "decode" else ("enc", "k_xf")
        if want in allk: (side, kname), kms = want, allk[want]
    # kms is the kernel's time per step; a kernel launched n times per step (k_xf: one launch per chunk of blocks of
    # the window-control pipeline) has kms = sum of its n launches, each timed by its own hipEvent pair on its stream.
    launches = float(enc.xf_launches()) if kname == "k_xf" else 1.0      # from the library: chunks of the last call
    # HBM bytes per launch of the roofline kernel: from the two PMC passes of THIS command (FETCH_SIZE / WRITE_SIZE cannot
    # be collected inside a timed run) - handed in with --pmc-summary (tools/gpu_r04.sh does the three runs on one box); null otherwise
    traffic, traffic_src = None, None
    # HBM bytes of the roofline kernel: the PMC counters cannot be collected inside a timed run, so the figure comes from the
    # counter summary of the SAME command (tools/pmc_summary.py over separate FETCH_SIZE / WRITE_SIZE passes): the one handed in
    # with --pmc-summary, else the committed profiles/r06_pmc_summary.json (or r05's) - and that one ONLY when it was taken on exactly the
    # sources this library is built from (`_meta.src_rev` == ulcx_build_rev()) and on this workload.  Anything else: null.
    lib_rev = ulc_amd.build_rev()
    cands = [args.pmc_summary] if args.pmc_summary else [os.path.join(ROOT, "profiles", f) for f in ("r06_pmc_summary.json", "r05_pmc_summary.json")]
    for pmc_path in cands:
        if not (pmc_path and os.path.exists(pmc_path)):
            continue
        try:
            js = json.load(open(pmc_path))
            meta = js.get("_meta", {})
            same_run = bool(args.pmc_summary)
            same_src = meta.get("src_rev") == lib_rev and lib_rev != "unknown"
            same_work = (meta.get("config") == args.config and meta.get("mode") == args.mode and meta.get("blocks") == K and meta.get("streams") == B and world == 1 and not args.pcm16)
            ent = js.get(kname, {})
            if traffic is None and ent.get("hbm_bytes_per_launch") is not None and (same_run or (same_src and same_work)):
                traffic = ent["hbm_bytes_per_launch"] / launches
                traffic_src = os.path.relpath(pmc_path, ROOT) + " (src_rev %s%s)" % (meta.get("src_rev", "?"), ", git %s" % meta["git"] if meta.get("git") else "")
        except Exception:
            traffic, traffic_src = None, None
    launch_bytes = alg_bytes_block * B * K / launches
    kms_launch = kms / launches
    achieved = launch_bytes / (kms_launch * 1e-3) / 1e9
    enc_total = sum(acc_enc.values()); dec_total = sum(acc_dec.values())
    leg_bytes = alg_bytes_block * B * K
    enc_gbs = leg_bytes / (enc_total * 1e-3) / 1e9 if enc_total else None
    dec_gbs = leg_bytes / (dec_total * 1e-3) / 1e9 if dec_total else None
    step_gbs = leg_bytes * len(legs) / (el_rank / args.steps) / 1e9         # (the step moves the algorithmic bytes once per leg)

    if rank == 0 and world == 1 and not args.no_cpu:
        S = 8
        sample = pcm[:S].cpu().numpy()
        cpu = cpu_baseline(sample, K, cfg, legs)

    # the other two configurations, short and outside the timed region: only on the driver's shape of run (headline config,
    # both legs, one GPU, the C API's f32)
    secondary = None
    if (rank == 0 and world == 1 and args.config == "vbr50" and args.mode == "both" and not args.pcm16 and not args.no_secondary
            and not args.streams and not args.blocks):
        secondary = {}
        for nm, (sb, sk) in {"cbr64_48k": (CONFIGS["cbr64_48k"]["total"] // 8, 16), "wswitch_4096": (CONFIGS["wswitch_4096"]["total"] // 8, 16)}.items():
            try:
                secondary[nm] = run_secondary(torch, ulc_amd, dev, nm, sb, sk)
            except Exception as e:                           # (never loses the headline line)
                secondary[nm] = {"error": repr(e)}

    if rank == 0:
        what = {"both": "encode+decode", "encode": "encode", "decode": "decode"}[args.mode]
        rc = "VBR -%g" % cfg["p0"] if cfg["mode"] == "vbr" else "CBR %g kbps" % cfg["p0"]
        wl = (f"{args.config} ({cfg['ref']}): Batch={B} independent {rate / 1000:g} kHz stereo streams/GPU x {K} blocks, BlockSize={bs}, {rc}; "
              f"step = {' then '.join(legs)}" + (f"; {total_streams} streams in total over {world} GPU(s)" if strong else "")
              + (" -- PCM16 ingest/output variant (int16 samples in HBM, not the C API's f32)" if args.pcm16 else ""))
        line = {
            "metric": f"{what} Msamples/s at BlockSize={bs} stereo (channel-samples through {what}, {rc}, {K} blocks per stream per call)",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" + (" (TEST RUN: ranks share one GPU, gloo control plane - not a result)" if share else "")
                    + (f" (TEST RUN: {total_streams} streams in total instead of the configuration's {cfg['total']} - not a result)" if strong and total_streams != cfg["total"] else ""),
            "config": {"workload": wl, "config": args.config, "mode": args.mode,
                       "streams_per_gpu": B, "blocks_per_stream_per_step": K, "block_size": bs, "channels": CH, "rate_hz": rate,
                       "parallelism": f"batch split over {world} GPU(s), no collective on the data path",
                       "per_rank_ms_per_step": [round(float(x), 4) for x in per_rank_ms]},
            "library_src_rev": lib_rev,
            "roofline": {"bound": "hbm", "kernel": f"{kname} ({side})", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_block": alg_bytes_block, "blocks_per_launch": B * K / launches,
                         "launches_per_step": launches, "kernel_ms": kms_launch,
                         # the pipeline, not only its largest kernel: every leg moves the same algorithmic bytes
                         "encode_achieved": enc_gbs, "encode_frac": enc_gbs / HBM_PEAK_GBS if enc_gbs else None,
                         "decode_achieved": dec_gbs, "decode_frac": dec_gbs / HBM_PEAK_GBS if dec_gbs else None,
                         "step_achieved": step_gbs, "step_frac": step_gbs / HBM_PEAK_GBS},
            "whole_pipeline": {"encode_ms": enc_total, "decode_ms": dec_total,
                               "encode_Msamples_s": B * K * bs * CH / (enc_total * 1e-3) / 1e6 if enc_total else None,
                               "decode_Msamples_s": B * K * bs * CH / (dec_total * 1e-3) / 1e6 if dec_total else None,
                               "mean_block_bytes": mean_bytes, "decode_ok": ok},
            "kernels_ms": {**{f"enc.{k_}": round(v, 4) for k_, v in acc_enc.items()}, **{f"dec.{k_}": round(v, 4) for k_, v in acc_dec.items()}},
        }
        if secondary is not None:
            line["secondary"] = secondary
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
