#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point rs_step (actions in, obs/reward/labels/violations out
every step) next to the resident path, for DESIGN.md §Measurement.  Not the bench metric."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
import numpy as np  # noqa: E402
from ranslice.config import make_config  # noqa: E402
from ranslice.fading import synth_fading  # noqa: E402
from ranslice.vec_env import VecRanSlice  # noqa: E402

N = 4096
env = VecRanSlice(n_envs=N, cfg=make_config(0, n_envs=N), fading=[synth_fading(t, 10000) for t in range(3)])
env.reset()
for i in range(300):
    env.random_actions(2024, i)
    env.step_resident()
acts = env.fetch()['actions']
rng = np.random.default_rng(0)
K = 200
t0 = time.perf_counter()
for i in range(K):
    env.step(acts)
t1 = time.perf_counter()
for i in range(K):
    env.random_actions(2024, 300 + i)
    env.step_resident()
env.synchronize()
t2 = time.perf_counter()
print('host-buffer rs_step : %.0f env-steps/s (%.3f ms/step)' % (N * K / (t1 - t0), 1e3 * (t1 - t0) / K))
print('resident path       : %.0f env-steps/s (%.3f ms/step)' % (N * K / (t2 - t1), 1e3 * (t2 - t1) / K))
