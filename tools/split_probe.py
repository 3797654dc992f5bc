#!/usr/bin/env python3
"""Config 3 (4096 replicas + one KBRL agent each) as P independent (simulator, agent) pairs of 4096 / P replicas on their own
HIP streams: the chains of different pairs are independent (replicas are), so the chip can run one pair's step kernel
(issue-bound) under another pair's agent kernels (latency-bound).  Same replica ids and seeds as the single pair, hence the
same trajectories.  usage: python tools/split_probe.py [--pairs 2] [--warmup 100] [--steps 200]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
import numpy as np  # noqa: E402
from ranslice.config import make_config, EMBB_A, EMBB_SEC  # noqa: E402
from ranslice.fading import synth_traces  # noqa: E402
from ranslice.kbrl_dev import VecKBRL  # noqa: E402
from ranslice.sharding import replica_seeds  # noqa: E402
from ranslice.vec_env import VecRanSlice  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--pairs', type=int, default=2)
ap.add_argument('--warmup', type=int, default=100)
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--profile', default='sos')
args = ap.parse_args()
N, P = args.envs, args.pairs
n = N // P
fading = synth_traces(10000, args.profile)
rng = np.random.default_rng(0)
ia = rng.integers(EMBB_A[0], EMBB_A[1], size=(N, 5)).astype(np.int32)
sf = rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(N, 5)).astype(np.int32)
pairs = []
for p in range(P):
    env = VecRanSlice(n_envs=n, cfg=make_config(0, n_envs=n), fading=fading)
    agent = VecKBRL(n, [10] * 5, 200, accuracy_range=(0.99, 0.999), capacity=4096, pool_bytes=(64 << 30) // P)
    env.reset(seeds=replica_seeds(0, p * n, n))
    agent.reset(ia[p * n:(p + 1) * n], sf[p * n:(p + 1) * n], seeds=replica_seeds(0, p * n, n))
    a0 = np.ascontiguousarray(ia[p * n:(p + 1) * n])
    env._check(env.L.rs_step(env.h, a0.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
    pairs.append((env, agent))


def run(k):
    for _ in range(k):
        for env, agent in pairs:
            agent.step_resident(env)
            env.step_resident()


def sync():
    for env, agent in pairs:
        env.synchronize()
        agent.synchronize()


run(args.warmup)
sync()
t0 = time.perf_counter()
run(args.steps)
sync()
dt = time.perf_counter() - t0
sizes = np.concatenate([a.dictionary_sizes() for _, a in pairs])
acts = np.concatenate([e.fetch()['actions'] for e, _ in pairs])
print(json.dumps({'pairs': P, 'envs': N, 'env_steps_per_s': N * args.steps / dt, 'ms_per_step': 1e3 * dt / args.steps,
                  'dictionary_size_mean': float(sizes.mean()), 'dictionary_size_max': int(sizes.max()),
                  'mean_action_sum': float(acts.sum(axis=1).mean())}))
