"""N>1 path on CPU: two gloo ranks shard a batch of independent streams exactly as bench.py
does across GPUs (contiguous split, no data-path collective), process their shards (the
oracle stands in for the device here — this is a test of the decomposition, not of the
kernels), and the concatenation equals the unsharded result; timing uses MAX over ranks."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import shard
    from ulc_testlib import synth_pcm, oracle_encode_stream
    lo, hi = shard.stream_range(total, rank, world)
    sizes = []
    for s in range(lo, hi):
        pcm = synth_pcm(s, 6 * 512, 2, 44100, transient=True, seed=1)
        out, bits, wc, cplx = oracle_encode_stream(pcm, 512, 44100, quality=50.0)
        sizes.append(int(bits.sum()))
    dist.barrier()
    t = shard.max_over_ranks(0.25 * (rank + 1), dist)                 # slowest rank defines the step
    # control-plane gather of per-stream byte counts (the "host concatenation" of SURVEY.md §8e)
    got = [None] * world
    dist.all_gather_object(got, (lo, hi, sizes))
    if rank == 0:
        q.put((t, got))
    dist.destroy_process_group()


def test_two_rank_shard_equals_unsharded():
    import shard
    from ulc_testlib import synth_pcm, oracle_encode_stream
    total, world = 7, 2                                                # uneven split on purpose
    assert [shard.stream_range(total, r, world) for r in range(world)] == [(0, 4), (4, 7)]
    assert list(shard.weak_scaling_ids(4096, 3))[:2] == [12288, 12289]
    assert shard.whole_job_throughput(100, 8, 2.0) == 400.0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs: p.start()
    t, got = q.get(timeout=120)
    for p in procs:
        p.join(60); assert p.exitcode == 0
    assert t == 0.5                                                    # MAX over ranks
    covered, sizes = [], []
    for lo, hi, sz in got:
        covered += list(range(lo, hi)); sizes += sz
    assert covered == list(range(total))
    ref = []
    for s in range(total):
        pcm = synth_pcm(s, 6 * 512, 2, 44100, transient=True, seed=1)
        ref.append(int(oracle_encode_stream(pcm, 512, 44100, quality=50.0)[1].sum()))
    assert sizes == ref
