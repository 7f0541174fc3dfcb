"""bench.py --gpus N starts its own N ranks (one process per GPU) when no launcher is around it.

CPU: on a box without GPUs `--gpus 2` must start two ranks, both must stop at the device check, the launcher must exit
non-zero and print no result line (never a silent `n_gpus: 1`).
GPU: on the 1-GPU box the two ranks share the device over a gloo control plane (ULCX_BENCH_SHARE_GPU=1: the N > 1 code
path of bench.py - sharding, barrier, MAX over ranks, gather of the per-rank times - with everything but RCCL's transport)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_without_devices_spawns_two_ranks_and_fails_cleanly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the run would succeed")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"])
    assert r.returncode != 0
    assert r.stderr.count("2 ranks but only") == 2, r.stderr          # both ranks started and both refused
    assert "no result line" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]   # nothing that could be read as a result


def test_launcher_and_world_size_must_agree():
    r = _run(["--gpus", "4"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in r.stderr


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_box():
    r = _run(["--gpus", "2", "--streams", "64", "--blocks", "4", "--steps", "2", "--warmup", "1", "--no-cpu"],
             env={"ULCX_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["config"]["per_rank_ms_per_step"]) == 2
    assert d["whole_pipeline"]["decode_ok"] is True
    assert "TEST RUN" in d["data"]
    # weak scaling: the whole job is both ranks' streams
    assert d["config"]["streams_per_gpu"] == 64
    assert abs(d["value"] - 2 * 64 * 4 * 2048 * 2 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6) < 1e-6 * d["value"]


def _one_line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_eight_ranks_share_the_gpu_box_weak_scaling():
    """World size 8 (the node the 1 -> 8 curve will be measured on) on the 1-GPU box: rendezvous of 8 ranks, 8 entries of
    per_rank_ms_per_step, MAX over ranks, whole-job value = 8 ranks' streams.  Eight small encoders share the device."""
    r = _run(["--gpus", "8", "--streams", "64", "--blocks", "4", "--steps", "2", "--warmup", "1", "--no-cpu"],
             env={"ULCX_BENCH_SHARE_GPU": "1"}, timeout=900)
    d = _one_line(r)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and len(d["config"]["per_rank_ms_per_step"]) == 8
    assert d["whole_pipeline"]["decode_ok"] is True and "TEST RUN" in d["data"]
    assert d["ms_per_step"] >= max(d["config"]["per_rank_ms_per_step"]) * 0.999       # the slowest rank defines the step
    assert abs(d["value"] - 8 * 64 * 4 * 2048 * 2 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]


@pytest.mark.gpu
@pytest.mark.parametrize("config,bs", [("cbr64_48k", 2048), ("wswitch_4096", 4096)])
def test_eight_ranks_strong_scaling_split_of_the_fixed_total_configurations(config, bs):
    """configs[3] / configs[4] are fixed totals split over the GPUs (shard.stream_range): 8 ranks, an uneven total (516 =
    4 x 65 + 4 x 64 streams), value = the TOTAL's samples over the slowest rank's time."""
    r = _run(["--gpus", "8", "--config", config, "--total-streams", "516", "--blocks", "4", "--steps", "2", "--warmup", "1", "--no-cpu"],
             env={"ULCX_BENCH_SHARE_GPU": "1"}, timeout=900)
    d = _one_line(r)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and len(d["config"]["per_rank_ms_per_step"]) == 8
    assert d["config"]["streams_per_gpu"] == 65 and "516 streams in total" in d["config"]["workload"]      # rank 0's share
    assert d["whole_pipeline"]["decode_ok"] is True and d["data"].count("TEST RUN") == 2
    assert abs(d["value"] - 516 * 4 * bs * 2 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_two_gpus_over_rccl():
    """The N > 1 path as the driver runs it: one process per GPU, torch.distributed over RCCL (bench.py: init_process_group
    "nccl").  Needs two visible GPUs: skipped on the 1-GPU box, the first real execution of that path on a multi-GPU one."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("fewer than two GPUs visible")
    r = _run(["--gpus", "2", "--streams", "256", "--blocks", "4", "--steps", "3", "--warmup", "1", "--no-cpu"], timeout=900)
    d = _one_line(r)
    assert d["n_gpus"] == 2 and len(d["config"]["per_rank_ms_per_step"]) == 2 and "TEST RUN" not in d["data"]
    assert d["whole_pipeline"]["decode_ok"] is True
    assert abs(d["value"] - 2 * 256 * 4 * 2048 * 2 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
