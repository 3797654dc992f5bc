"""Host logic of the shared-dictionary exchange on CPU: the merge is sharding-independent, and the
torch.distributed exchange (all_gather; RCCL on GPUs, gloo here) delivers every rank the same merged list."""
import os
import socket

import numpy as np

from ranslice.kbrl_dev import PROP_W, merge_proposals


def _fake_props(rng, ids, budget, S=2):
    counts = np.zeros(S, dtype=np.int32)
    props = np.zeros((S, budget, PROP_W))
    for s in range(S):
        mine = [i for i in ids if rng.random() < 0.6]
        counts[s] = len(mine)
        for j, i in enumerate(mine[:budget]):
            props[s, j, 0] = i
            props[s, j, 1] = 4 * (i % 50) + (i % 2)
            props[s, j, 2:] = i * 0.001 + np.arange(PROP_W - 2)
    return counts, props


def test_merge_is_sharding_independent():
    budget = 6
    whole_ids = list(range(40))
    for trial in range(20):
        seeds = np.random.default_rng(trial)
        # per-replica "has a proposal" decided by replica id only, so every sharding sees the same proposers
        flags = {s: {i for i in whole_ids if np.random.default_rng([trial, s, i]).random() < 0.5} for s in range(2)}

        def build(ids):
            counts = np.zeros(2, dtype=np.int32)
            props = np.zeros((2, budget, PROP_W))
            for s in range(2):
                mine = [i for i in ids if i in flags[s]]
                counts[s] = len(mine)
                for j, i in enumerate(mine[:budget]):
                    props[s, j, 0] = i
                    props[s, j, 2] = i * 0.5
            return counts, props
        c1, p1 = build(whole_ids)
        m1 = merge_proposals(c1[None], p1[None], budget)
        for W in (2, 4, 5):
            per = 40 // W
            cs, ps = zip(*[build(whole_ids[w * per:(w + 1) * per]) for w in range(W)])
            mW = merge_proposals(np.stack(cs), np.stack(ps), budget)
            assert (m1[0] == mW[0]).all() and m1[1].tobytes() == mW[1].tobytes()
            assert (mW[2].sum(axis=0) == mW[0]).all()
            # accepted proposers of a rank are a prefix of its list
            for w in range(W):
                for s in range(2):
                    assert mW[2][w, s] <= min(cs[w][s], budget)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # tests/dist_util.py (torch.distributed helpers of the gloo tests)
    from dist_util import gather_exchange
    from ranslice.kbrl_dev import merge_proposals
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    counts, props = _fake_props(rng, range(rank * 10, rank * 10 + 10), budget=4)
    all_c, all_p, me = gather_exchange(device='cpu')(counts, props)
    mc, mp, taken = merge_proposals(all_c, all_p, 4)
    q.put((rank, me, mc.tolist(), float(mp.sum()), taken.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_gloo():
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
    (r0, me0, c0, s0, t0), (r1, me1, c1, s1, t1) = res
    assert (me0, me1) == (0, 1)
    assert c0 == c1 and s0 == s1 and t0 == t1      # both ranks hold the same merged list
    assert all(x <= 4 for x in c0)
