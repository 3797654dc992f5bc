#!/usr/bin/env python3
"""Why is embb_step slower in the KBRL closed loop?  (a) random actions, (b) KBRL actions, (c) KBRL kernels run but
their actions are overwritten by random ones (same launch pattern as (b), same work as (a))."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
import numpy as np
from ranslice.config import make_config
from ranslice.fading import synth_fading
from ranslice.kbrl_dev import VecKBRL
from ranslice.vec_env import VecRanSlice
N = 4096
fading = [synth_fading(t, 10000) for t in range(3)]
for mode in ('random', 'kbrl', 'kbrl-kernels+random-actions'):
    env = VecRanSlice(n_envs=N, cfg=make_config(0, n_envs=N), fading=fading)
    env.reset()
    agent = VecKBRL(N, [10] * 5, 200, capacity=256)
    rng = np.random.default_rng(0)
    ia = rng.integers(4, 20, size=(N, 5)).astype(np.int32)
    agent.reset(ia, rng.integers(2, 8, size=(N, 5)).astype(np.int32))
    env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
    def adv(i):
        if mode != 'random':
            agent.step_resident(env)
        if mode != 'kbrl':
            env.random_actions(2024, i)
        env.step_resident()
    for i in range(300):
        adv(i)
    env.synchronize(); agent.synchronize()
    env.set_kernel_timing(True)
    c0 = env.counters()
    t0 = time.perf_counter()
    for i in range(100):
        adv(300 + i)
    env.synchronize(); agent.synchronize()
    dt = time.perf_counter() - t0
    ms, n = env.kernel_time_ms()
    c1 = env.counters()
    out = env.fetch()
    print('%-30s embb %.3f ms  step %.3f ms  UEs/slice %.2f  PF iters/env-step %.0f  samples/env-step %.0f  mean PRBs %.1f' % (
        mode, ms, 1e3 * dt / 100, (c1[3] - c0[3]) / (100 * N * 5 * 50), (c1[2] - c0[2]) / (100 * N), (c1[0] - c0[0]) / (100 * N),
        out['actions'].sum(axis=1).mean()), flush=True)
    env.close(); agent.close()
