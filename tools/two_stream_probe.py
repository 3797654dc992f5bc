#!/usr/bin/env python3
"""Probe: 4096 replicas as K handles of 4096/K on their own streams (the tail of one sub-batch's step kernel overlaps the
body of the next): env-steps/s against the single handle.  Same replicas, same seeds, same results (sharding-invariant)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
from ranslice.config import make_config  # noqa: E402
from ranslice.fading import synth_fading  # noqa: E402
from ranslice.sharding import replica_seeds  # noqa: E402
from ranslice.vec_env import VecRanSlice  # noqa: E402

N = 4096
fading = [synth_fading(t, 10000) for t in range(3)]
for K in (1, 2, 4):
    n = N // K
    envs = []
    for k in range(K):
        e = VecRanSlice(n_envs=n, cfg=make_config(0, n_envs=n), fading=fading)
        e.set_group_size(16)
        e.reset(seeds=replica_seeds(0, k * n, n))
        envs.append(e)
    for i in range(1500):
        for e in envs:
            e.random_actions(2024, i)
            e.step_resident()
    for e in envs:
        e.synchronize()
    t0 = time.perf_counter()
    S = 300
    for i in range(S):
        for e in envs:
            e.random_actions(2024, 1500 + i)
            e.step_resident()
    for e in envs:
        e.synchronize()
    dt = time.perf_counter() - t0
    print('%d handle(s) x %d replicas: %.3f ms per step of all %d, %.2f M env-steps/s' % (K, n, 1e3 * dt / S, N, N * S / dt / 1e6), flush=True)
    for e in envs:
        e.close()
