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


_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def replica_seed(base_seed, replica_id):
    """64-bit stream seed of global replica `replica_id` of the batch seeded with `base_seed`: a SplitMix64 mix of
    both, so that batches with nearby base seeds share no trajectories (base + id would make replica 1 of seed 0 the
    same environment as replica 0 of seed 1)."""
    return _splitmix64((_splitmix64(int(base_seed) & _M64) + int(replica_id)) & _M64)


def replica_seeds(base_seed, first, count):
    """Seeds of the global replicas first .. first+count-1, whatever the number of GPUs: a replica's trajectory
    does not depend on how the batch is sharded (tests/test_gpu_parity.py)."""
    return np.array([replica_seed(base_seed, i) for i in range(first, first + count)], dtype=np.uint64)


def aggregate_throughput(units_per_rank, world, seconds_max):
    """whole-job units/s: every rank processed units_per_rank in at most seconds_max"""
    return units_per_rank * world / seconds_max
