"""Replica sharding across the GPUs of one node.

The reference's only parallelism is one OS process per independent run (reference
experiments_kbrl.py:69-70): env replicas never exchange data.  The build keeps that: rank r of W
owns a contiguous range of global replica ids, every rank holds its own copy of the read-only
fading/MCS tables, and RanSlice.step needs NO collective.  (The measurement harness agrees on wall-clock
time through a process group of its own: tools/dist_util.py.)
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


def aggregate_throughput(units_per_rank, world, seconds_max):
    """whole-job units/s: every rank processed units_per_rank in at most seconds_max"""
    return units_per_rank * world / seconds_max
