"""N>1 path on CPU: two gloo processes shard the replica range exactly as bench.py does on RCCL and
agree on the max-over-ranks time.  (The simulator itself needs a GPU; what is covered here is the
host logic that makes the multi-GPU run correct by construction: disjoint complete shards, seeds
that do not depend on the sharding, MAX/SUM reductions.)"""
import os
import socket

import numpy as np
import pytest

from ranslice.sharding import shard_range, replica_seeds


def test_shard_range_properties():
    for n in (0, 1, 7, 4096, 65536, 65537):
        for w in (1, 2, 3, 4, 8):
            got = []
            for r in range(w):
                first, cnt = shard_range(n, r, w)
                got.extend(range(first, first + cnt))
                assert abs(cnt - n / w) < 1
            assert got == list(range(n))
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_seeds_independent_of_sharding():
    whole = replica_seeds(5, 0, 64)
    parts = np.concatenate([replica_seeds(5, *shard_range(64, r, 4)) for r in range(4)])
    assert (whole == parts).all() and whole.dtype == np.uint64


def _worker(rank, world, port, q):
    import torch.distributed as dist
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # tests/dist_util.py (torch.distributed helpers of the gloo tests)
    from dist_util import max_over_ranks, sum_over_ranks
    from ranslice.sharding import shard_range, replica_seeds, aggregate_throughput
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    first, cnt = shard_range(8192, rank, world)
    seeds = replica_seeds(0, first, cnt)
    elapsed = 1.0 + 0.5 * rank          # rank 1 is the slow one
    tmax = max_over_ranks(elapsed)
    tot = sum_over_ranks([cnt, int(seeds.sum() % 1000003)])
    dist.barrier()
    q.put((rank, first, cnt, tmax, tot.tolist(), aggregate_throughput(cnt * 10, world, tmax)))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, f0, c0, t0, s0, v0), (r1, f1, c1, t1, s1, v1) = res
    assert (f0, c0, f1, c1) == (0, 4096, 4096, 4096)
    assert t0 == t1 == 1.5                      # MAX over ranks
    assert s0 == s1 and s0[0] == 8192           # SUM over ranks
    assert v0 == v1 == 4096 * 10 * 2 / 1.5
