"""Replica sharding across the GPUs of one node.

The reference's only parallelism is one OS process per independent run (reference
experiments_kbrl.py:69-70): env replicas never exchange data.  The build keeps that: rank r of W
owns a contiguous range of global replica ids, every rank holds its own copy of the read-only
fading/MCS tables, and RanSlice.step needs NO collective.  torch.distributed (RCCL on GPUs, gloo in
the CPU tests) is used only to agree on wall-clock time and to gather scalar reports.
"""
import numpy as np


def shard_range(global_n, rank, world):
    """Contiguous, balanced split of range(global_n): returns (first, count) of this rank."""
    if not (0 <= rank < world) or global_n < 0:
        raise ValueError('bad shard request')
    base, extra = divmod(global_n, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def replica_seeds(base_seed, first, count):
    """Stream seed of global replica i is base_seed + i, whatever the number of GPUs: a replica's
    trajectory does not depend on how the batch is sharded (tests/test_gpu_parity.py)."""
    return (np.uint64(base_seed) + np.arange(first, first + count, dtype=np.uint64)).astype(np.uint64)


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the timed region ends when the slowest rank ends)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device=None):
    """SUM all-reduce of a small vector of counters."""
    import torch
    import torch.distributed as dist
    v = np.asarray(values, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return v
    t = torch.tensor(v, dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def aggregate_throughput(units_per_rank, world, seconds_max):
    """whole-job units/s: every rank processed units_per_rank in at most seconds_max"""
    return units_per_rank * world / seconds_max
