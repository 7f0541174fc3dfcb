"""Multi-GPU decomposition of the hot path: streams are independent, so the
batch splits contiguously across ranks and NO collective sits on the data path.  The only
cross-rank operations are control plane: a barrier around the timed region and a MAX over
ranks of the elapsed time (bench.py).  Backend-agnostic (RCCL on GPUs, gloo in CPU tests)."""


def stream_range(total_streams, rank, world):
    """Contiguous shard [lo, hi) of the global stream index range owned by `rank`."""
    base, rem = divmod(total_streams, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def weak_scaling_ids(streams_per_rank, rank):
    """Global stream ids of a rank when per-rank work is fixed (bench.py: scaling = weak)."""
    return range(rank * streams_per_rank, (rank + 1) * streams_per_rank)


def max_over_ranks(value, dist=None, device=None):
    """MAX over ranks of a python float (the slowest rank defines the step time)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_throughput(units_per_rank, world, seconds):
    """Aggregate rate over all ranks: every rank processed units_per_rank in `seconds` (max over ranks)."""
    return units_per_rank * world / seconds
